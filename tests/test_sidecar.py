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
