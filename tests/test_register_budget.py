"""CPU (hipcc cross-compiles): the two kernels of the default two-level step must fit one CU TOGETHER -- the matrix-core-bound
panel kernel (8 waves, two per SIMD) beside a gather-bound window workgroup (one wave per SIMD), 512 registers per SIMD lane
and 160 KiB of LDS per CU (profiles/HISTORY.md section 3.3).  The fit is exact today (2 x 176 + 160 = 512): one more register in either
kernel and the pair silently stops overlapping, so the compiler's own resource report is asserted here."""
import os
import re
import subprocess

from conftest import REPO

SOURCE = r'''
#include "voltrix/spmm_kernels.hpp"
#include "voltrix/spmm_panel_kernels.hpp"
template __global__ void voltrix::spmm_tc16_pair_kernel<voltrix::SpmmTile<128, 3, 4, 2, false, false>>(
    const voltrix::SpmmArgs<voltrix::SpmmTile<128, 3, 4, 2, false, false>>);
template __global__ void voltrix::spmm_tc16_kernel<voltrix::SpmmTile<128, 3, 4, 2, false, false>>(
    const voltrix::SpmmArgs<voltrix::SpmmTile<128, 3, 4, 2, false, false>>);
template __global__ void voltrix::spmm_panel_kernel<voltrix::PanelTile<128, 3, 8, 4, 1, false>>(
    const voltrix::PanelArgs<voltrix::PanelTile<128, 3, 8, 4, 1, false>>);
static_assert(voltrix::SpmmTile<128, 3, 4, 2, false, false>::BLOCK_LDS + voltrix::PanelTile<128, 3, 8, 4, 1, false>::BLOCK_LDS
                  <= 160 * 1024, "one window workgroup + one panel workgroup per CU");
'''


def _usage(text):
    out = {}
    for block in text.split("remark: Function Name: ")[1:]:
        name = block.split(" ")[0]
        grab = lambda key: int(re.search(key + r": (\d+)", block).group(1))  # noqa: E731
        out[name] = {"vgpr": grab("VGPRs"), "agpr": grab("AGPRs"), "scratch": grab(r"ScratchSize \[bytes/lane\]"),
                     "occupancy": grab(r"Occupancy \[waves/SIMD\]")}
    return out


def test_panel_and_window_kernels_share_a_cu(tmp_path):
    src = tmp_path / "pair.hip"
    src.write_text(SOURCE)
    inc = os.path.join(REPO, "voltrix-spmm_amd", "voltrix", "include")
    run = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", inc,
                          "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", str(tmp_path / "pair.o")],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    usage = _usage(run.stderr)
    pair = next(v for k, v in usage.items() if "spmm_tc16_pair_kernel" in k)
    single = next(v for k, v in usage.items() if "spmm_tc16_kernel" in k)
    panel = next(v for k, v in usage.items() if "spmm_panel_kernel" in k)
    assert pair["scratch"] == single["scratch"] == panel["scratch"] == 0           # no spills anywhere on the hot path
    granule = lambda r: (r + 7) // 8 * 8  # noqa: E731
    panel_regs = granule(panel["vgpr"] + panel["agpr"])
    for name, k in (("pair", pair), ("single", single)):
        regs = granule(k["vgpr"] + k["agpr"])
        assert 2 * panel_regs + regs <= 512, (name, panel, k)      # two panel waves + one window wave per SIMD
    # LDS: the static_assert in SOURCE (window ring 4 x 3 x 8 KiB + metadata, panel ring 24 KiB + metadata <= 160 KiB)
