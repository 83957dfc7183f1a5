"""Ahead-of-time population of the JIT cache (hipcc cross-compiles gfx950 without a GPU).

``__graft_entry__.build()`` calls this in the build container with ``VOLTRIX_CACHE_DIR`` pointing in-tree, so the
kernel directories travel to the GPU box and the first ``voltrix.spmm`` there is a cache hit, not a compile.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import torch

from ..jit import build, cpp_format, generate
from . import bmat_swizzle, hmat_gem, preprocess, spmm


def _spmm_arg_defs(dtype):
    return spmm.arg_defs_for(dtype)


def jobs(feature_widths=(32, 64, 128), modes=("default", "none", "stream")):
    """(name, arg_defs, code) of every kernel the operator API can ask for with the current VOLTRIX_TUNE_SPACE."""
    out = []
    for mod in (preprocess, hmat_gem, bmat_swizzle):
        name = {preprocess: "preprocess_kernel", hmat_gem: "hmat_gen_kernel",
                bmat_swizzle: "hmat_packed_swizzle_kernel"}[mod]
        out.append((name, mod.arg_defs, generate(mod.includes, mod.arg_defs, mod.template)))
    seen = set()
    for dtype, eb in ((torch.float16, 2), (torch.bfloat16, 2), (torch.float32, 4)):
        for width in feature_widths:
            points = []
            saved = os.environ.get("VOLTRIX_TUNE_SPACE")
            for mode in modes:  # the tuned space and the single default tile (VOLTRIX_TUNE_SPACE=none)
                os.environ["VOLTRIX_TUNE_SPACE"] = mode
                points += list(spmm.tile_space(width, eb, dtype == torch.bfloat16))
            if saved is None:
                os.environ.pop("VOLTRIX_TUNE_SPACE", None)
            else:
                os.environ["VOLTRIX_TUNE_SPACE"] = saved
            if eb == 2:   # weighted SpMM (voltrix/weighted.py): the default tile of every width; other points build on demand
                os.environ["VOLTRIX_TUNE_SPACE"] = "none"
                points += list(spmm.tile_space(width, eb, dtype == torch.bfloat16, weighted=True))
                if saved is None:
                    os.environ.pop("VOLTRIX_TUNE_SPACE", None)
                else:
                    os.environ["VOLTRIX_TUNE_SPACE"] = saved
            for point in points:
                key = (eb, point["BF16"], point["FS"], point["DEPTH"], point["WAVES"], point["SCHED"], point["WEIGHTED"])
                if key in seen:
                    continue
                seen.add(key)
                arg_defs = _spmm_arg_defs(dtype)
                out.append(("spmm_kernel", arg_defs, generate(spmm.includes, arg_defs, cpp_format(spmm.template, point))))
    return out


def prebuild(feature_widths=(32, 64, 128), workers=None, prune=False) -> int:
    """``prune``: afterwards remove every other ``kernel.*`` directory of the cache (entries of older header versions --
    the cache key hashes the include tree -- which would only travel to the GPU box as dead weight)."""
    todo = jobs(feature_widths)
    workers = workers or max(1, min(len(todo), os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        runtimes = list(pool.map(lambda j: build(*j), todo))
    if prune and runtimes:
        import shutil

        keep = {os.path.realpath(r.path) for r in runtimes}
        root = os.path.dirname(next(iter(keep)))
        for entry in os.listdir(root):
            full = os.path.realpath(os.path.join(root, entry))
            if entry.startswith("kernel.") and os.path.isdir(full) and full not in keep:
                shutil.rmtree(full, ignore_errors=True)
    return len(todo)
