"""Registry of the acceleration side-cars ``csr_preprocess`` builds beside a reference handle (round 4).

``voltrix.csr_preprocess`` must return the reference's three tensors, byte for byte; the two-level form of the same matrix
(voltrix/hybrid.py) can therefore only ride along.  Rounds 2-3 hung it on the ``hspa_packed`` tensor OBJECT as a Python
attribute: any new Python object over the same memory (a view, ``.detach()``, a tuple re-packed through ``torch.Tensor``
methods) silently lost it, and the operator fell back to the window format (1.35 -> 1.91 ms on the headline graph) without a
word.  Now the side-car is keyed by the MEMORY the handle lives in:

    key = (device, storage address, storage offset of the tensor, number of elements)

so every tensor object that aliases the handle's ``hspa_packed`` finds it.  The entry dies with the storage (a finalizer on
the storage object -- torch keeps one Python object per storage), so an address reused by a later allocation can never
inherit a stale side-car.  What still cannot be followed is a COPY of the bytes (``.clone()``, ``.to(device)``, pickling, a
handle rebuilt from saved tensors): for those there is ``copy_side_car(src, dst)`` and the ``save_handle`` / ``load_handle``
pair, and ``voltrix.spmm`` warns -- once per process -- when a handle big enough to matter reaches it without any record of a
decision (neither a side-car nor "decided: window format").
"""
from __future__ import annotations

import dataclasses
import threading
import warnings
import weakref
from typing import Optional

import torch

# re-entrant: the storage finalizer (_drop) may run inside a garbage-collection pass that fires while this thread holds the
# lock in lookup() / register()
_LOCK = threading.RLock()
_ENTRIES = {}          # key -> TwoLevelHandle, or None = "csr_preprocess decided for the window format"
_SLIM = set()          # keys of stand-in handles (slim_handle): nothing but the side-car is left of them
_WARNED = [False]


def _key(t: torch.Tensor):
    st = t.untyped_storage()
    return (t.device.type, t.device.index, st.data_ptr(), t.storage_offset(), t.numel()), st


def register(hspa_packed: torch.Tensor, two) -> None:
    """Record the decision for this handle: ``two`` (a ``hybrid.TwoLevelHandle``) or None (window format)."""
    key, storage = _key(hspa_packed)
    with _LOCK:
        fresh = key not in _ENTRIES
        _ENTRIES[key] = two
    if fresh:
        weakref.finalize(storage, _drop, key)


def _drop(key) -> None:
    with _LOCK:
        _ENTRIES.pop(key, None)
        _SLIM.discard(key)
        _CSR.pop(key, None)


# ---- CSR side-car (round 6): handles of short windows keep the device CSR they were built from, for the CSR row-gather kernel
# ---- (spmm_csr_kernels.hpp); voltrix.spmm times it once per (width, dtype) against the block-format path and keeps the faster
_CSR = {}


class CsrSideCar:
    """Device CSR of a handle (int32 ``indptr`` [N + 1], ``indices`` [nnz], duplicate-free) + what ``voltrix.spmm`` decided per
    (padded width, dtype): "csr" | "block"."""
    __slots__ = ("indptr", "indices", "num_rows", "num_cols", "choice")

    def __init__(self, indptr, indices, num_rows, num_cols):
        self.indptr, self.indices, self.num_rows, self.num_cols, self.choice = indptr, indices, num_rows, num_cols, {}


def register_csr(hspa_packed: torch.Tensor, csr: "CsrSideCar") -> None:
    key, storage = _key(hspa_packed)
    with _LOCK:
        fresh = key not in _ENTRIES and key not in _CSR
        _CSR[key] = csr
    if fresh:
        weakref.finalize(storage, _drop, key)


def lookup_both(hspa_packed: torch.Tensor):
    """``(known, two, csr)`` with ONE key computation (``voltrix.spmm`` asks for both on every call: the key costs a storage lookup)."""
    key, _ = _key(hspa_packed)
    with _LOCK:
        if key in _ENTRIES:
            return True, _ENTRIES[key], _CSR.get(key)
        return False, None, _CSR.get(key)


def lookup_csr(hspa_packed: torch.Tensor):
    key, _ = _key(hspa_packed)
    with _LOCK:
        return _CSR.get(key)


def lookup(hspa_packed: torch.Tensor):
    """``(known, two)``: ``known`` is False when nothing was ever recorded for this memory (a copy, a reloaded handle)."""
    key, _ = _key(hspa_packed)
    with _LOCK:
        if key in _ENTRIES:
            return True, _ENTRIES[key]
    return False, None


def copy_side_car(src_hspa_packed: torch.Tensor, dst_hspa_packed: torch.Tensor) -> bool:
    """After copying a handle's tensors (``.clone()``, ``.to(device)``): let the copy use the original's side-car.  The
    side-car's own tensors stay where they are -- same device only.  Returns whether ``src`` had a record."""
    known, two = lookup(src_hspa_packed)
    if known:
        assert two is None or two.hspa_packed.device == dst_hspa_packed.device, "the side-car lives on another device"
        register(dst_hspa_packed, two)
    return known


def warn_if_unknown(hspa_packed: torch.Tensor, num_nodes: int, num_edges: int, min_edges: int, min_rows: int,
                    min_mean_degree: float) -> None:
    """``voltrix.spmm`` in auto mode, no record for this memory: say so once when the graph is of the size and density for
    which ``csr_preprocess`` would have considered the two-level form."""
    if _WARNED[0] or num_edges < min_edges or num_nodes < min_rows or num_edges < min_mean_degree * max(1, num_nodes):
        return
    _WARNED[0] = True
    warnings.warn(
        "voltrix.spmm: this handle's hspa_packed did not come from csr_preprocess in this process (a clone, a .to(), a "
        "reloaded file?), so the two-level side-car, if one was built, is not attached to it and the product runs in the "
        "window format.  Use voltrix.copy_side_car(original_hspa_packed, copy) after copying a handle, or "
        "voltrix.save_handle / voltrix.load_handle to move one between processes.", stacklevel=3)


def slim_handle(handle):
    """Opt-in, for hosts short of memory: once ``csr_preprocess`` has decided for the two-level form, the reference handle of
    the WHOLE matrix is only the key the side-car hangs on -- ``voltrix.spmm`` never reads its TC blocks again (617 MB on the
    headline graph next to a 703 MB side-car).  Returns ``(blk_offsets, stub, stub)`` with 4-element stand-ins for
    ``hspa_packed`` and ``hind`` that carry the side-car (and the hash tag); drop the original tuple to free the memory.
    What is given up: the window-format paths of THIS handle -- ``VOLTRIX_HYBRID=0``, ``VOLTRIX_FP32_MODE=exact``, a direct
    ``spmm_kernel`` call, the C-ABI -- which ``voltrix.spmm`` then refuses instead of running on four elements.  A handle
    without a side-car is returned unchanged."""
    blk_offsets, hspa_packed, hind = handle
    known, two = lookup(hspa_packed)
    if not known or two is None:
        return handle
    stub_packed = torch.zeros(4, dtype=hspa_packed.dtype, device=hspa_packed.device)
    stub_hind = torch.zeros(4, dtype=hind.dtype, device=hind.device)
    if getattr(hspa_packed, "hash_tag", None) is not None:
        stub_packed.hash_tag = hspa_packed.hash_tag
    register(stub_packed, two)
    with _LOCK:
        _SLIM.add(_key(stub_packed)[0])
    return blk_offsets, stub_packed, stub_hind


def is_slim(hspa_packed: torch.Tensor) -> bool:
    """Whether this memory is a stand-in made by ``slim_handle`` (the window-format paths must refuse it)."""
    key, _ = _key(hspa_packed)
    with _LOCK:
        return key in _SLIM


# ---- moving a handle (and its side-car) between processes ---------------------------------------------------------------

def save_handle(path: str, handle, num_nodes: int) -> None:
    """``torch.save`` of the three reference tensors plus whatever ``csr_preprocess`` attached to them (the two-level
    side-car: residual handle, panel plan, stage records of the one-launch form) -- everything on the CPU."""
    blk_offsets, hspa_packed, hind = handle
    known, two = lookup(hspa_packed)
    blob = {"format": "voltrix-handle-1", "num_nodes": int(num_nodes),
            "handle": [t.detach().cpu() for t in (blk_offsets, hspa_packed.view(torch.int32), hind)],
            "hash_tag": getattr(hspa_packed, "hash_tag", None), "decided": bool(known), "two_level": None}
    if two is not None:
        plan = two.plan
        blob["two_level"] = {
            "residual": [t.detach().cpu() for t in (two.blk_offsets, two.hspa_packed.view(torch.int32), two.hind)],
            "plan_tensors": {k: (getattr(plan, k).detach().cpu().view(torch.int32) if getattr(plan, k) is not None else None)
                             for k in ("panel_ptr", "panel_cols", "panel_bits", "panel_order", "xcd_ptr")},
            "parts_cap": plan.parts.cap if plan.parts is not None else None,   # the part table is rebuilt from this on load
            "plan_scalars": {f.name: getattr(plan, f.name) for f in dataclasses.fields(plan)
                             if isinstance(getattr(plan, f.name), (int, float, str, bool))
                             or f.name in ("num_nodes", "waves", "row_blocks", "tau", "num_ksteps", "num_shared_edges",
                                           "num_resid_edges", "max_panels_per_xcd")},
            "window_xcd_ptr": two.window_xcd_ptr.cpu() if two.window_xcd_ptr is not None else None,
            "num_nodes": two.num_nodes, "num_edges": two.num_edges, "hash_tag": two.hash_tag,
            "format_choice": dict(two.format_choice),
            "fused": None if two.fused is None else {"wave_ptr": two.fused.wave_ptr.cpu(),
                                                     "records": two.fused.records.view(torch.int32).cpu(),
                                                     "num_records": two.fused.num_records},
        }
    torch.save(blob, path)


def load_handle(path: str, device: Optional[torch.device] = None):
    """-> ``(blk_offsets, hspa_packed, hind)`` on ``device`` (default: the current CUDA device), side-car re-attached."""
    from . import hybrid

    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    # the blob is tensors in plain containers (save_handle): the restricted unpickler is enough, a handle file from another host
    # cannot run code here
    blob = torch.load(path, map_location="cpu", weights_only=True)
    assert isinstance(blob, dict), f"{path}: not a voltrix handle file"
    assert blob.get("format") == "voltrix-handle-1", f"{path}: not a voltrix handle file"
    blk_offsets, packed, hind = (t.to(device) for t in blob["handle"])
    hspa_packed = packed.view(torch.uint32)
    if blob.get("hash_tag") is not None:
        hspa_packed.hash_tag = blob["hash_tag"]
    tl = blob["two_level"]
    two = None
    if tl is not None:
        pt = {k: (v.to(device) if v is not None else None) for k, v in tl["plan_tensors"].items()}
        pt["panel_bits"] = pt["panel_bits"].view(torch.uint32)
        scalars = {k: tl["plan_scalars"][k] for k in ("num_nodes", "waves", "row_blocks", "tau", "num_ksteps",
                                                       "num_shared_edges", "num_resid_edges", "max_panels_per_xcd")}
        plan = hybrid.PanelPlan(**pt, **scalars)
        if tl.get("parts_cap") is not None:
            plan.parts = hybrid.panel_parts(plan.panel_ptr, tl["parts_cap"], plan.xcd_ptr)
        r0, r1, r2 = (t.to(device) for t in tl["residual"])
        two = hybrid.TwoLevelHandle(r0, r1.view(torch.uint32), r2, plan, tl["num_nodes"], tl["num_edges"],
                                    hash_tag=tl["hash_tag"], format_choice=dict(tl["format_choice"]))
        if tl.get("window_xcd_ptr") is not None:
            two.window_xcd_ptr = tl["window_xcd_ptr"].to(device)
        if tl["fused"] is not None:
            two.fused = hybrid.FusedRecords(tl["fused"]["wave_ptr"].to(device),
                                            tl["fused"]["records"].to(device).view(torch.uint32), tl["fused"]["num_records"])
    if blob["decided"]:
        register(hspa_packed, two)
    return blk_offsets, hspa_packed, hind
