"""hipcc driver + on-disk kernel cache.

Counterpart of the reference's voltrix/jit/compiler.py (nvcc, sm_90a): same cache layout
``$VOLTRIX_CACHE_DIR | ~/.voltrix-spmm`` ``/cache/kernel.<name>.<md5[:12]>/`` + ``/tmp``, same atomic
``os.replace`` publication (compiler.py:109-114,185), same kind of signature (kernel name, hash of the device
headers, generated code, compiler identity, flags -- compiler.py:140-142).  The compiler is hipcc for
``--offload-arch=gfx950``; it cross-compiles without a GPU, so ``__graft_entry__.build()`` can pre-populate
an in-tree cache that travels to the GPU box.
"""
from __future__ import annotations

import functools
import hashlib
import os
import re
import subprocess
import uuid
from typing import Tuple

from ..project import (
    CACHE_DIR_FLAG,
    DEBUG_FLAG,
    HIPCC_COMPILER_FLAG,
    JIT_PRINT_NVCC_COMMAND_FLAG,
    NVCC_COMPILER_FLAG,
    PROJECT_NAME_ABBR_LOWER,
    PROJECT_NAME_FULL_LOWER,
    PTXAS_VERBOSE_FLAG,
)
from .runtime import Runtime, RuntimeCache
from .template import args_to_text

runtime_cache = RuntimeCache()


def hash_to_hex(s: str) -> str:
    """First 12 hex digits of md5 (reference compiler.py:25-28; golden-tested)."""
    return hashlib.md5(s.encode("utf-8")).hexdigest()[0:12]


@functools.lru_cache(maxsize=None)
def get_jit_include_dir() -> str:
    return os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include"))


@functools.lru_cache(maxsize=None)
def jit_header_closure() -> Tuple[str, ...]:
    """The device headers a JIT-compiled kernel can see: the ``"voltrix/*.hpp"`` entries named by the kernel modules
    (``jit_kernels/*.py``: ``includes = (...)``) and everything they include, transitively.  Headers of the ahead-of-time
    library only (plan builders, unit tables, the search of reorder_kernels.hpp ...) are not in it."""
    root = get_jit_include_dir()
    modules = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jit_kernels"))
    pattern = re.compile(r'"(%s/[\w./]+\.(?:hpp|h))"' % PROJECT_NAME_ABBR_LOWER)
    todo = []
    for fn in sorted(os.listdir(modules)):
        if fn.endswith(".py"):
            with open(os.path.join(modules, fn)) as f:
                todo += pattern.findall(f.read())
    seen = []
    while todo:
        rel = todo.pop()
        if rel in seen:
            continue
        path = os.path.join(root, rel)
        assert os.path.isfile(path), f"Cannot find device header {path}"
        seen.append(rel)
        with open(path) as f:
            todo += re.findall(r'#include\s+"(%s/[\w./]+)"' % PROJECT_NAME_ABBR_LOWER, f.read())
    return tuple(sorted(seen))


@functools.lru_cache(maxsize=None)
def get_repo_version() -> str:
    """md5 over every device header a JIT kernel can include: editing a kernel invalidates the cache (reference
    compiler.py:45-59 hashes its whole include tree; here the tree also holds the ahead-of-time library's headers, which no
    JIT kernel sees, so the hash follows the include graph instead -- ``jit_header_closure``)."""
    root = get_jit_include_dir()
    assert os.path.isdir(os.path.join(root, PROJECT_NAME_ABBR_LOWER)), f"Cannot find include directory {root}"
    md5 = hashlib.md5()
    for rel in jit_header_closure():
        with open(os.path.join(root, rel), "rb") as f:
            md5.update(f.read())
    return md5.hexdigest()[0:12]


@functools.lru_cache(maxsize=None)
def get_kernel_sources_version() -> str:
    """md5 over EVERY device source of the package -- all headers under ``include/voltrix`` (JIT-reachable or not: the panel,
    plan and table kernels live in the ahead-of-time library) and the library's ``csrc/*``.  Measurements that cannot be
    repeated live (the PMC passes behind ``profiles/traffic.json``) carry it, and ``bench.py`` replays them only while it
    matches the tree (round 5)."""
    root = get_jit_include_dir()
    files = [os.path.join(root, PROJECT_NAME_ABBR_LOWER, f) for f in sorted(os.listdir(os.path.join(root, PROJECT_NAME_ABBR_LOWER)))]
    csrc = os.path.normpath(os.path.join(root, "..", "..", "csrc"))
    if os.path.isdir(csrc):
        files += [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".hpp", ".h"))]
    md5 = hashlib.md5()
    for path in files:
        if os.path.isfile(path):
            with open(path, "rb") as f:
                md5.update(f.read())
    return md5.hexdigest()[0:12]


@functools.lru_cache(maxsize=None)
def get_hipcc_compiler() -> Tuple[str, str]:
    """(path, version) of the first usable hipcc: $VOLTRIX_HIPCC_COMPILER, $VOLTRIX_NVCC_COMPILER (legacy name),
    $ROCM_PATH/bin/hipcc, /opt/rocm/bin/hipcc (reference get_nvcc_compiler, compiler.py:62-81)."""
    candidates = [os.getenv(HIPCC_COMPILER_FLAG), os.getenv(NVCC_COMPILER_FLAG)]
    if os.getenv("ROCM_PATH"):
        candidates.append(os.path.join(os.environ["ROCM_PATH"], "bin", "hipcc"))
    candidates.append("/opt/rocm/bin/hipcc")
    pattern = re.compile(r"HIP version:\s*([\d.]+)")
    for path in candidates:
        if path and os.path.exists(path):
            # the version WITHOUT starting a program (round 6): a cached kernel must load without any spawn -- `hipcc --version`
            # runs hipconfig, and a process whose GPU is already initialised (every program under `rocprofv3 --pmc`) must not
            # exec anything on this pool.  <rocm>/include/hip/hip_version.h holds the same "major.minor.patch" hipcc prints.
            version = _hip_version_from_header(path)
            if version is None:
                out = subprocess.run([path, "--version"], capture_output=True, text=True).stdout
                match = pattern.search(out)
                assert match, f"Cannot get the version of HIP compiler {path}"
                version = match.group(1)
            return path, version
    raise RuntimeError("Cannot find any available hipcc compiler")


def _hip_version_from_header(hipcc_path: str):
    """"7.2.26015" from the hip_version.h that ships beside ``hipcc`` (None when it is not there)."""
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc_path)))
    header = os.path.join(root, "include", "hip", "hip_version.h")
    try:
        with open(header) as f:
            text = f.read()
    except OSError:
        return None
    parts = [re.search(rf"#define\s+HIP_VERSION_{k}\s+(\d+)", text) for k in ("MAJOR", "MINOR", "PATCH")]
    return ".".join(m.group(1) for m in parts) if all(parts) else None


get_nvcc_compiler = get_hipcc_compiler  # drop-in alias (reference name)


def get_offload_arch() -> str:
    return "gfx950"   # the one target of this package (MI355X / CDNA4)


def get_default_user_dir() -> str:
    if CACHE_DIR_FLAG in os.environ:
        path = os.environ[CACHE_DIR_FLAG]
        os.makedirs(path, exist_ok=True)
        return path
    return os.path.join(os.path.expanduser("~"), f".{PROJECT_NAME_FULL_LOWER}")


def get_tmp_dir() -> str:
    return os.path.join(get_default_user_dir(), "tmp")


def get_cache_dir() -> str:
    return os.path.join(get_default_user_dir(), "cache")


def make_tmp_dir() -> str:
    os.makedirs(get_tmp_dir(), exist_ok=True)
    return get_tmp_dir()


def put(path: str, data, is_binary: bool = False) -> None:
    """Write then POSIX-atomic replace, so concurrent builders never expose a torn file."""
    tmp = os.path.join(make_tmp_dir(), f"file.tmp.{uuid.uuid4()}.{hash_to_hex(path)}")
    with open(tmp, "wb" if is_binary else "w") as f:
        f.write(data)
    os.replace(tmp, path)


def compile_flags() -> list:
    flags = [
        "-std=c++17",
        "-shared",
        "-fPIC",
        "-O3",
        f"--offload-arch={get_offload_arch()}",
        "-Wno-unused-function",
        "-Wno-deprecated-declarations",
    ]
    if PTXAS_VERBOSE_FLAG in os.environ:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    return flags


def kernel_dir(name: str, code: str) -> str:
    flags = compile_flags()
    signature = f"{name}$${get_repo_version()}$${code}$${get_hipcc_compiler()}$${flags}"
    return os.path.join(get_cache_dir(), f"kernel.{name}.{hash_to_hex(signature)}")


# what this process has done: kernels found in the on-disk / in-memory cache vs. kernels it had to compile (tests, bench.py)
build_stats = {"cache_hits": 0, "compiled": 0}


def build(name: str, arg_defs: tuple, code: str) -> Runtime:
    path = kernel_dir(name, code)
    cached = runtime_cache[path]
    build_stats["cache_hits" if cached is not None else "compiled"] += 1
    if cached is not None:
        if os.getenv(DEBUG_FLAG, None):
            print(f"Using cached JIT runtime {os.path.basename(path)} during build")
        return cached

    os.makedirs(path, exist_ok=True)
    src_path = os.path.join(path, "kernel.hip")
    put(os.path.join(path, "kernel.args"), args_to_text(arg_defs))
    put(src_path, code)

    so_path = os.path.join(path, "kernel.so")
    tmp_so = os.path.join(make_tmp_dir(), f"hipcc.tmp.{uuid.uuid4()}.{hash_to_hex(so_path)}.so")
    command = [get_hipcc_compiler()[0], src_path, "-o", tmp_so, *compile_flags(), f"-I{get_jit_include_dir()}"]
    if os.getenv(DEBUG_FLAG, None) or os.getenv(JIT_PRINT_NVCC_COMMAND_FLAG, False):
        print(f"Compiling JIT runtime {os.path.basename(path)} with command {command}")
    proc = subprocess.run(command, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"Failed to compile {src_path}:\n{proc.stderr[-4000:]}")
    if PTXAS_VERBOSE_FLAG in os.environ:
        print(proc.stderr)
    os.replace(tmp_so, so_path)

    runtime_cache[path] = Runtime(path)
    return runtime_cache[path]
