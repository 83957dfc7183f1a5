"""CPU: the side-car registry (voltrix/sidecar.py) -- keyed by the MEMORY of ``hspa_packed``: views and re-packed tuples find
the record, copies do not, a freed storage takes its record with it (an address reused by a later tensor never inherits a
stale side-car), the warning for unknown big handles fires once.  GPU behaviour: tests/test_gpu_hybrid.py."""
import gc
import warnings

import torch

from voltrix import sidecar


class _Two:          # stands in for hybrid.TwoLevelHandle: the registry never looks inside
    hspa_packed = torch.zeros(1)


def test_views_find_the_record_copies_do_not_and_it_dies_with_the_storage():
    t = torch.arange(64, dtype=torch.int32)
    two = _Two()
    assert sidecar.lookup(t) == (False, None)
    sidecar.register(t, two)
    assert sidecar.lookup(t) == (True, two)
    assert sidecar.lookup(t.view(-1))[1] is two and sidecar.lookup(t.detach())[1] is two
    assert sidecar.lookup(tuple([t])[0])[1] is two
    assert sidecar.lookup(t.view(torch.int32).view(4, 16).view(-1))[1] is two
    assert sidecar.lookup(t[4:]) == (False, None)             # another extent of the same storage is not the handle
    assert sidecar.lookup(t.clone()) == (False, None)
    c = t.clone()
    assert sidecar.copy_side_car(t, c) and sidecar.lookup(c)[1] is two
    assert not sidecar.copy_side_car(torch.zeros(3), c)
    sidecar.register(t, None)                                  # a later decision replaces the record
    assert sidecar.lookup(t) == (True, None)
    key = sidecar._key(t)[0]
    assert key in sidecar._ENTRIES
    del t
    gc.collect()
    assert key not in sidecar._ENTRIES                         # the finalizer of the storage dropped it


def test_unknown_big_handle_warns_once():
    sidecar._WARNED[0] = False
    t = torch.zeros(8, dtype=torch.int32)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        sidecar.warn_if_unknown(t, 1000, 10, 1 << 22, 131072, 64)             # small graph: silent
        assert not caught
        sidecar.warn_if_unknown(t, 300000, 1 << 25, 1 << 22, 131072, 64)      # big, dense, unknown: once
        sidecar.warn_if_unknown(t, 300000, 1 << 25, 1 << 22, 131072, 64)
    assert len(caught) == 1 and "copy_side_car" in str(caught[0].message)
    sidecar._WARNED[0] = False


def test_slim_handle_keeps_the_side_car_and_marks_the_stand_ins():
    """voltrix.slim_handle (round 4, opt-in): a handle with a side-car is replaced by 4-element stand-ins that carry it (and
    the tag); a handle without one is returned as it is; the mark dies with the stand-in's storage."""
    import gc

    from voltrix import sidecar

    blk = torch.zeros(3, dtype=torch.int32)
    packed, hind = torch.zeros(400, dtype=torch.int32), torch.zeros(800, dtype=torch.int32)
    assert sidecar.slim_handle((blk, packed, hind)) == (blk, packed, hind)          # nothing recorded: unchanged
    sidecar.register(packed, None)
    assert sidecar.slim_handle((blk, packed, hind))[1] is packed                   # decided: window format -> unchanged
    marker = object()
    sidecar.register(packed, marker)
    packed.hash_tag = "tagged"
    b2, p2, h2 = sidecar.slim_handle((blk, packed, hind))
    assert b2 is blk and p2.numel() == 4 and h2.numel() == 4 and p2.dtype == packed.dtype and h2.dtype == hind.dtype
    assert sidecar.lookup(p2) == (True, marker) and p2.hash_tag == "tagged"
    assert sidecar.is_slim(p2) and sidecar.is_slim(p2.view(torch.int32)) and not sidecar.is_slim(packed)
    key = sidecar._key(p2)[0]
    del p2, b2, h2
    gc.collect()
    assert key not in sidecar._SLIM and key not in sidecar._ENTRIES
