"""CPU (hipcc cross-compiles): the stream kernel's ring is reused on a RUN-TIME vmcnt budget that counts exactly ``ns_live``
``global_store_dwordx4`` per finished unit, issued after the step's LDS-DMA loads (spmm_stream_kernels.hpp, store_unit / wait_vm).
A store that the compiler merged, split or hoisted above the loads would make the wait return before a stage has landed --
silently wrong sums.  The stores are inline asm since round 6; this reads the disassembly."""
import os
import re
import subprocess

import pytest

from conftest import REPO

SOURCE = r'''
#include "voltrix/spmm_stream_kernels.hpp"
using T = voltrix::SpmmTile<%d, 2, 2, 2, false, false>;
template __global__ void voltrix::spmm_stream_kernel<T, false>(const voltrix::StreamArgs<T>);
'''


@pytest.mark.parametrize("fs", [128, 32])
def test_stream_kernel_stores_are_counted_instructions_after_the_loads(tmp_path, fs):
    src = tmp_path / "stream.hip"
    src.write_text(SOURCE % fs)
    inc = os.path.join(REPO, "voltrix-spmm_amd", "voltrix", "include")
    out = tmp_path / "stream.s"
    run = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", inc, "-S", "--cuda-device-only",
                          str(src), "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = out.read_text().splitlines()
    start = next(i for i, line in enumerate(lines) if re.match(r"^_ZN7voltrix\S*spmm_stream_kernel\S*:", line))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = [line.split(";")[0].strip() for line in lines[start:end]]
    stores = [(i, t) for i, t in enumerate(body) if t.startswith("global_store")]
    dmas = [i for i, t in enumerate(body) if t.startswith("global_load_lds_dwordx4")]
    slots = fs // 16
    # one instruction per 16-column slot in each of the four forms of store_unit (full / partial slab x scaled / unscaled): nothing
    # merged (dwordx4 is the widest), nothing split, nothing shared between the forms
    assert all(t.startswith("global_store_dwordx4 ") for _, t in stores), [t for _, t in stores if "dwordx4" not in t]
    assert len(stores) == 4 * slots, len(stores)
    offsets = sorted(int(re.search(r"offset:(\w+)", t).group(1), 0) if "offset:" in t else 0 for _, t in stores)
    assert offsets == sorted(4 * [64 * s for s in range(slots)])
    # ... and all of them after the loop's row gathers (the prologue's gathers come earlier still)
    assert dmas and min(i for i, _ in stores) > max(dmas)
    # no other global memory instruction can slip into the counted window: atomics / plain loads appear only before the loop
    late = [t for i, t in enumerate(body) if i > max(dmas) and re.match(r"global_(load|atomic)", t)]
    assert not late, late
